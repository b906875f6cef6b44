// Shared helpers for liblasr (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>
#include <hip/hip_bf16.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include <string.h>

#include "../../include/lasr.h"

namespace lasr {

// ---- error state (thread local, message only; codes travel as return values) -----------------
void set_error(const char* fmt, ...);
int fail(int code, const char* fmt, ...);
int hip_fail(hipError_t e, const char* what);

#define LASR_CHECK_ARG(cond, ...)                         \
  do {                                                    \
    if (!(cond)) return ::lasr::fail(LASR_E_ARG, __VA_ARGS__); \
  } while (0)
#define LASR_CHECK_SHAPE(cond, ...)                         \
  do {                                                      \
    if (!(cond)) return ::lasr::fail(LASR_E_SHAPE, __VA_ARGS__); \
  } while (0)
#define LASR_LAUNCH_CHECK(name)                                   \
  do {                                                            \
    hipError_t e__ = hipGetLastError();                           \
    if (e__ != hipSuccess) return ::lasr::hip_fail(e__, name);    \
  } while (0)
#define LASR_TRY(expr)          \
  do {                          \
    int rc__ = (expr);          \
    if (rc__ != 0) return rc__; \
  } while (0)

// roctx ranges (capi.hip): named host-side ranges around the plan's stages and units for `rocprofv3 --marker-trace`; off unless
// LASR_ROCTX=1 (librocprofiler-sdk-roctx.so / libroctx64.so is dlopen'd on first use - a missing library switches the ranges off)
void roctx_push(const char* name);
void roctx_pop();
struct RoctxRange {
  explicit RoctxRange(const char* name) { roctx_push(name); }
  ~RoctxRange() { roctx_pop(); }
  RoctxRange(const RoctxRange&) = delete;
};

// in-library kernel timer (capi.hip); kinds are LASR_PROF_* in lasr.h
int prof_begin(int kind, hipStream_t st, double flops, double bytes);
void prof_end(int token, hipStream_t st);

static inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }
static inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }
static inline size_t dtype_size(int dtype) { return dtype == LASR_BF16 ? 2 : 4; }

// ---- bf16 <-> f32 -------------------------------------------------------------------------
typedef uint16_t bf16_t;  // raw storage

__device__ __forceinline__ float bf16_to_f32(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
__device__ __forceinline__ bf16_t f32_to_bf16(float f) {
  // plain cast: hipcc emits v_cvt_pk_bf16_f32 (RNE, NaN stays NaN)
  __hip_bfloat16 h = __float2bfloat16(f);
  return *reinterpret_cast<bf16_t*>(&h);
}

template <typename T>
struct Elem;
template <>
struct Elem<float> {
  static constexpr int kDtype = LASR_F32;
  static constexpr int kVec = 4;  // elements per 16-byte access
  __device__ static __forceinline__ float ld(const float* p) { return *p; }
  __device__ static __forceinline__ void st(float* p, float v) { *p = v; }
  __device__ static __forceinline__ void ld4(const float* p, float (&o)[4]) {
    float4 v = *reinterpret_cast<const float4*>(p);
    o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
  }
  __device__ static __forceinline__ void st4(float* p, const float (&o)[4]) {
    *reinterpret_cast<float4*>(p) = make_float4(o[0], o[1], o[2], o[3]);
  }
};
template <>
struct Elem<bf16_t> {
  static constexpr int kDtype = LASR_BF16;
  static constexpr int kVec = 8;
  __device__ static __forceinline__ float ld(const bf16_t* p) { return bf16_to_f32(*p); }
  __device__ static __forceinline__ void st(bf16_t* p, float v) { *p = f32_to_bf16(v); }
  __device__ static __forceinline__ void ld4(const bf16_t* p, float (&o)[4]) {
    uint2 v = *reinterpret_cast<const uint2*>(p);
    o[0] = __uint_as_float(v.x << 16); o[1] = __uint_as_float(v.x & 0xffff0000u);
    o[2] = __uint_as_float(v.y << 16); o[3] = __uint_as_float(v.y & 0xffff0000u);
  }
  __device__ static __forceinline__ void st4(bf16_t* p, const float (&o)[4]) {
    uint2 v;
    v.x = (uint32_t)f32_to_bf16(o[0]) | ((uint32_t)f32_to_bf16(o[1]) << 16);
    v.y = (uint32_t)f32_to_bf16(o[2]) | ((uint32_t)f32_to_bf16(o[3]) << 16);
    *reinterpret_cast<uint2*>(p) = v;
  }
};

// 16-byte vectors of activations: 4 f32 or 8 bf16 per lane per access
template <typename T>
struct Vec;
template <>
struct Vec<float> {
  static constexpr int kN = 4;
  __device__ static __forceinline__ void load(const float* p, float (&o)[4]) { Elem<float>::ld4(p, o); }
  // raw 16 bytes now, element values later (keeps batched loads at 4 registers each)
  __device__ static __forceinline__ uint4 raw(const float* p) { return *reinterpret_cast<const uint4*>(p); }
  __device__ static __forceinline__ void unpack(const uint4& v, float (&o)[4]) {
    o[0] = __uint_as_float(v.x); o[1] = __uint_as_float(v.y); o[2] = __uint_as_float(v.z); o[3] = __uint_as_float(v.w);
  }
  __device__ static __forceinline__ void store(float* p, const float (&o)[4]) { Elem<float>::st4(p, o); }
};
template <>
struct Vec<bf16_t> {
  static constexpr int kN = 8;
  __device__ static __forceinline__ uint4 raw(const bf16_t* p) { return *reinterpret_cast<const uint4*>(p); }
  // 16 bytes with the non-temporal hint (global_load_dwordx4 ... nt): for tensors read ONCE, long after they were written (the forward
  // tensors the backward re-reads).  Measured on this machine (tools/micro/hbm_bw.hip, profiles/r04_hbm_bw.txt): a 96 MB tensor read
  // back after 1.5 GB of other read + write traffic takes 32.5 us with default loads and 21.8 us with nt loads - a default load
  // allocates its line in the memory-side cache, which by then is full of dirty lines that have to be written back first.
  template <bool NTL>
  __device__ static __forceinline__ uint4 raw_if_nt(const bf16_t* p) {
    if constexpr (NTL) {
      typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
      const u32x4_t v = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(p));
      return make_uint4(v.x, v.y, v.z, v.w);
    } else {
      return *reinterpret_cast<const uint4*>(p);
    }
  }
  __device__ static __forceinline__ void unpack(const uint4& v, float (&o)[8]) {
    o[0] = __uint_as_float(v.x << 16); o[1] = __uint_as_float(v.x & 0xffff0000u);
    o[2] = __uint_as_float(v.y << 16); o[3] = __uint_as_float(v.y & 0xffff0000u);
    o[4] = __uint_as_float(v.z << 16); o[5] = __uint_as_float(v.z & 0xffff0000u);
    o[6] = __uint_as_float(v.w << 16); o[7] = __uint_as_float(v.w & 0xffff0000u);
  }
  __device__ static __forceinline__ void load(const bf16_t* p, float (&o)[8]) { unpack(raw(p), o); }
  __device__ static __forceinline__ uint4 pack(const float (&o)[8]) {
    uint4 v;
    v.x = (uint32_t)f32_to_bf16(o[0]) | ((uint32_t)f32_to_bf16(o[1]) << 16);
    v.y = (uint32_t)f32_to_bf16(o[2]) | ((uint32_t)f32_to_bf16(o[3]) << 16);
    v.z = (uint32_t)f32_to_bf16(o[4]) | ((uint32_t)f32_to_bf16(o[5]) << 16);
    v.w = (uint32_t)f32_to_bf16(o[6]) | ((uint32_t)f32_to_bf16(o[7]) << 16);
    return v;
  }
  __device__ static __forceinline__ void store(bf16_t* p, const float (&o)[8]) { *reinterpret_cast<uint4*>(p) = pack(o); }
};

// out[i] = sum_p partials[p][i] for i < ncols, in a fixed order; i < split goes to out0, the rest to
// out1[i - split] (out1 may be null).  One launch, 32 columns x 8 partial lanes per block (norm.hip).
int launch_reduce_partials(const float* partials, int n_part, int64_t ncols, float* out0, int64_t split, float* out1,
                           hipStream_t st);

// LASR_NT_LOADS: bit mask of the backward kernels that read their COLD operands (forward tensors saved for backward) with
// non-temporal loads: 1 = BN backward statistics (y, y2), 2 = BN backward apply (y, y2), 4 = depthwise backward (x),
// 8 = stage-batched 1x1 weight gradients (dy, dy2, u, x).  Default: see nt_loads_mask() in capi.hip (measured per site).
int nt_loads_mask();

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains vmcnt: every outstanding GLOBAL load and store
// of the wave must retire before the barrier (workgroup-scope release of global memory).  Inside a per-time-step recurrence
// whose global loads are prefetched steps ahead and whose stores are read by later kernels only, that drain IS the step
// time (measured: 0.65 us per LSTM step with it).  Use only where the waves exchange data through LDS alone.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// se.hip: back through the excite MLP of a ContextSE unit, for the whole batch, in two launches (se_bwd_hidden_kernel,
// se_bwd_pool_kernel).  Either from ds = d(loss)/d(scale) [B][C] (lasr_se_bwd), or - norm.hip's fused SE + BN backward - from the
// per-(utterance, slab) raw sums of bn_bwd_stats(per_utt), in which case the second launch also folds the BN-backward constants
// `tab` and the BN parameter gradients.  work: se_bwd_work_bytes(B, C).
struct SeBwdBn {            // the BN side of the fused form (partials == nullptr: not used)
  const float* partials; int nslab;                      // [B * nslab][4][C] raw sums
  const float* gamma; const float* beta; const float* ysum;
  const float* coef; const float* saved; const float* coef2; const float* saved2; const float* gamma2;
  float inv_n; float* tab; float* dgamma; float* dbeta; float* dgamma2; float* dbeta2;
};
size_t se_bwd_work_bytes(int64_t B, int64_t C);
int launch_se_bwd(const float* ds, const SeBwdBn* bn, const float* scale, const float* hidden, const float* pooled, const float* W1,
                  const float* W2, int64_t B, int64_t T_, int64_t C, float* seg, float* dW1, float* dW2, void* work, hipStream_t st);

// pass 2a: per-channel constants of the backward apply, folded once by C threads:
//   dy = G*d1 + Bc*y + Cc   with G = gamma*rstd, Bc = -G*rstd*s2/n, Cc = G*(mean*rstd*s2/n - s1/n)
// tab = [a1 | b1 | G1 | B1 | C1 | a2 | b2 | G2 | B2 | C2][C]; also emits dgamma = s2, dbeta = s1.
__device__ __forceinline__ void bn_bwd_table_channel(int c, int C, float s1, float s2, float s1b, float s2b, bool has2,
                                                     const float* __restrict__ coef, const float* __restrict__ saved,
                                                     const float* __restrict__ gamma, const float* __restrict__ coef2,
                                                     const float* __restrict__ saved2, const float* __restrict__ gamma2, float inv_n,
                                                     float* __restrict__ tab, float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                     float* __restrict__ dgamma2, float* __restrict__ dbeta2) {
  {
    const float q = saved[C + c], w = s2 * inv_n, G = gamma[c] * q;
    tab[c] = coef[c]; tab[C + c] = coef[C + c]; tab[2 * C + c] = G; tab[3 * C + c] = -G * q * w;
    tab[4 * C + c] = G * (saved[c] * q * w - s1 * inv_n);
    if (dbeta) dbeta[c] = s1;
    if (dgamma) dgamma[c] = s2;
  }
  if (has2) {
    const float q = saved2[C + c], w = s2b * inv_n, G = gamma2[c] * q;
    tab[5 * C + c] = coef2[c]; tab[6 * C + c] = coef2[C + c]; tab[7 * C + c] = G; tab[8 * C + c] = -G * q * w;
    tab[9 * C + c] = G * (saved2[c] * q * w - s1b * inv_n);
    if (dbeta2) dbeta2[c] = s1b;
    if (dgamma2) dgamma2[c] = s2b;
  } else {
    for (int k = 5; k < 10; ++k) tab[k * C + c] = 0.f;
  }
}


// ---- wave / block reductions (wave = 64 lanes) -------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

__device__ __forceinline__ float act_fwd(float x, int act) {
  if (act == LASR_ACT_RELU) return fmaxf(x, 0.f);
  if (act == LASR_ACT_SWISH) return x / (1.f + __expf(-x));
  return x;
}

// A runtime activation code as a compile-time constant inside f: f(std::integral_constant<int, ACT>).  With `act` a plain kernel
// argument, act_fwd / act_grad compile to two or three scalar compare-and-branch pairs PER ELEMENT in the BN passes' inner loops
// (128 branches in the statistics pass's loop body, read off the ISA in round 4) - every element its own basic block, nothing
// scheduled across them.  One uniform branch per kernel instead.
template <typename F>
__device__ __forceinline__ void with_act(int act, F&& f) {
  if (act == LASR_ACT_RELU) f(std::integral_constant<int, LASR_ACT_RELU>{});
  else if (act == LASR_ACT_SWISH) f(std::integral_constant<int, LASR_ACT_SWISH>{});
  else f(std::integral_constant<int, LASR_ACT_NONE>{});
}

}  // namespace lasr
