// Dev experiment (round-3 groundwork, not part of liblasr): the stride-1 depthwise conv of a 512-channel unit on TIME-MAJOR tensors,
// x_tm / u_tm [B][C][Tp] bf16 (Tp = T' rounded up, pad frames zero), against the channels-last product kernel's 13.8 us.
// Same banded-Toeplitz MFMA (v_mfma_f32_16x16x32_bf16, A built from packed tap tables) as csrc/conv.hip, but
//   * the channel rows are COPIED into the LDS image (16-byte chunks, no [frame][channel] staging, no ds_read_tr transposition),
//   * the accumulators (lane = 16-frame block x 4 consecutive frames, one channel per MFMA) are stored straight to HBM: a wave
//     instruction writes 512 contiguous bytes of one channel row - no output image.
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -o build/dw_tm tools/micro/dw_tm.hip && build/dw_tm
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <vector>

typedef unsigned short bf16_t;
typedef __bf16 dw_bf16x8 __attribute__((ext_vector_type(8)));
typedef float dw_f32x4 __attribute__((ext_vector_type(4)));

static inline bf16_t f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fffu + ((u >> 16) & 1u); return (bf16_t)(u >> 16); }
static inline float bf2f(bf16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }
__device__ __forceinline__ uint32_t d_f2bf(float f) { uint32_t u = __float_as_uint(f); u += 0x7fffu + ((u >> 16) & 1u); return u >> 16; }

constexpr int kCB = 64, TT = 512, KWMAX = 128, TIN = TT - 16 + KWMAX, LDI = 1296, WROW = 160;
constexpr int IMG_BYTES = kCB * LDI, SMEM = IMG_BYTES + kCB * WROW * 4;   // 82 944 + 40 960

template <int NKS, int VAR>   // VAR 0: full; 1: tap tables from a constant (no weight loads); 2: no output stores; 3: no tile loads
__global__ __launch_bounds__(512, 1) void dw_tm_kernel(const bf16_t* __restrict__ x, const float* __restrict__ w, bf16_t* __restrict__ y,
                                                       int T, int Tp, int C, int k) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  char* img = smem_raw;
  uint32_t* wsm = reinterpret_cast<uint32_t*>(smem_raw + IMG_BYTES);
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int b = blockIdx.y, c0 = blockIdx.x * kCB;
  {   // packed tap pair tables (csrc/conv.hip)
    constexpr int kIt = kCB * WROW / 512;
    float f0[kIt], f1[kIt];
#pragma unroll
    for (int it = 0; it < kIt; ++it) {
      const int i = tid + 512 * it;
      const int ch = i / WROW, idx = i - ch * WROW;
      const int i0 = idx < 80 ? 2 * idx : 2 * (idx - 80) + 1;
      const int j0 = i0 - 24, j1 = i0 - 23;
      const float* wc = w + (size_t)(c0 + ch) * k;
      const int q0 = min(max(j0, 0), k - 1), q1 = min(max(j1, 0), k - 1);
      const float a0 = VAR == 1 ? 0.01f * q0 : wc[q0], a1 = VAR == 1 ? 0.01f * q1 : wc[q1];
      f0[it] = (j0 >= 0 && j0 < k) ? a0 : 0.f;
      f1[it] = (j1 >= 0 && j1 < k) ? a1 : 0.f;
    }
#pragma unroll
    for (int it = 0; it < kIt; ++it) wsm[tid + 512 * it] = d_f2bf(f0[it]) | (d_f2bf(f1[it]) << 16);
  }
  const int pad = k / 2, P = (pad + 7) & ~7, sh = P - pad;
  constexpr int KW = 32 * NKS;
  const int n16 = lane & 15, g4 = lane >> 4;
  for (int tA = 0; tA < T; tA += TT) {
    const int tin = TT - 16 + KW;
    const int ncr = tin / 8;                                    // 16-byte chunks per channel row
    // ---- phase 1: the 64 channel rows, 16 bytes per lane, consecutive lanes along a row -------------------------------------
    constexpr int kMaxIt = (kCB * (TIN / 8) + 511) / 512;         // 10
    uint4 v[kMaxIt];
#pragma unroll
    for (int it = 0; it < kMaxIt; ++it) {
      const int i = tid + 512 * it;
      const int ch = min(i / ncr, kCB - 1), q = i - (i / ncr) * ncr;
      const int t0 = tA - P + 8 * q;
      const bool ok = i < kCB * ncr && t0 >= 0 && t0 + 8 <= Tp;
      const uint4 ld = VAR == 3 ? make_uint4(i, q, t0, ch) : *reinterpret_cast<const uint4*>(x + ((size_t)b * C + c0 + ch) * Tp + min(max(t0, 0), Tp - 8));
      const uint32_t mk = ok ? 0xffffffffu : 0u;
      v[it] = make_uint4(ld.x & mk, ld.y & mk, ld.z & mk, ld.w & mk);
    }
    __syncthreads();                                              // the previous tile's readers are done
#pragma unroll
    for (int it = 0; it < kMaxIt; ++it) {
      const int i = tid + 512 * it;
      if (i < kCB * ncr) {
        const int ch = i / ncr, q = i - ch * ncr;
        *reinterpret_cast<uint4*>(img + ch * LDI + q * 16) = v[it];
      }
    }
    __syncthreads();
    // ---- phase 2: as csrc/conv.hip ---------------------------------------------------------------------------------------------
    dw_f32x4 acc[8][2];
#pragma unroll
    for (int ch = 0; ch < 8; ++ch) {
      const int cl = wid * 8 + ch;
      acc[ch][0] = dw_f32x4{0.f, 0.f, 0.f, 0.f}; acc[ch][1] = dw_f32x4{0.f, 0.f, 0.f, 0.f};
      const char* row = img + cl * LDI;
      const int s0 = 8 * g4 - n16 - sh + 24;
      const uint32_t* wrow = wsm + cl * WROW + ((s0 & 1) ? 80 + ((s0 - 1) >> 1) : (s0 >> 1));
      uint32_t wa[NKS][4];
      uint4 b0[NKS], b1[NKS];
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) {
#pragma unroll
        for (int i = 0; i < 4; ++i) wa[ks][i] = wrow[16 * ks + i];
        b0[ks] = *reinterpret_cast<const uint4*>(row + (16 * n16 + 32 * ks + 8 * g4) * 2);
        b1[ks] = *reinterpret_cast<const uint4*>(row + (256 + 16 * n16 + 32 * ks + 8 * g4) * 2);
      }
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) {
        union { uint32_t u[4]; dw_bf16x8 v; } af, bf0, bf1;
#pragma unroll
        for (int i = 0; i < 4; ++i) af.u[i] = wa[ks][i];
        bf0.u[0] = b0[ks].x; bf0.u[1] = b0[ks].y; bf0.u[2] = b0[ks].z; bf0.u[3] = b0[ks].w;
        bf1.u[0] = b1[ks].x; bf1.u[1] = b1[ks].y; bf1.u[2] = b1[ks].z; bf1.u[3] = b1[ks].w;
        acc[ch][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af.v, bf0.v, acc[ch][0], 0, 0, 0);
        acc[ch][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af.v, bf1.v, acc[ch][1], 0, 0, 0);
      }
    }
    // ---- phase 3: straight from the accumulators: lane = (block n16, frames 4*g4 .. +3), one channel per MFMA -------------------
#pragma unroll
    for (int ch = 0; ch < 8; ++ch) {
      bf16_t* yr = y + ((size_t)b * C + c0 + wid * 8 + ch) * Tp;
#pragma unroll
      for (int ns = 0; ns < 2; ++ns) {
        const int t = tA + ns * 256 + 16 * n16 + 4 * g4;
        uint32_t e[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) e[r] = t + r < T ? d_f2bf(acc[ch][ns][r]) : 0u;      // pad frames stay zero
        if (VAR == 2 ? (e[0] == 0x12345u && t < 0) : t < Tp) *reinterpret_cast<uint2*>(yr + t) = make_uint2(e[0] | (e[1] << 16), e[2] | (e[3] << 16));
      }
    }
  }
}

int main() {
  const int B = 32, C = 512, T = 501, Tp = 512, k = 63;
  const size_t n = (size_t)B * C * Tp;
  std::vector<bf16_t> hx(n, 0), hy(n);
  std::vector<float> hw((size_t)C * k);
  srand(1);
  for (int b = 0; b < B; ++b) for (int c = 0; c < C; ++c) for (int t = 0; t < T; ++t) hx[((size_t)b * C + c) * Tp + t] = f2bf((rand() % 2001 - 1000) / 1000.f);
  for (auto& v : hw) v = bf2f(f2bf((rand() % 2001 - 1000) / 8000.f));
  bf16_t *dx, *dy; float* dw;
  (void)hipMalloc(&dx, n * 2); (void)hipMalloc(&dy, n * 2); (void)hipMalloc(&dw, hw.size() * 4);
  (void)hipMemcpy(dx, hx.data(), n * 2, hipMemcpyHostToDevice); (void)hipMemcpy(dw, hw.data(), hw.size() * 4, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  float tot = 0; const int reps = 20;
  auto run = [&](int var) {
    float t = 0;
    for (int r = 0; r < reps + 2; ++r) {
      (void)hipEventRecord(e0, 0);
#define LAUNCH(V_) do { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(dw_tm_kernel<3, V_>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
      hipLaunchKernelGGL((dw_tm_kernel<3, V_>), dim3(C / kCB, B), dim3(512), SMEM, 0, dx, dw, dy, T, Tp, C, k); } while (0)
      if (var == 0) LAUNCH(0); else if (var == 1) LAUNCH(1); else if (var == 2) LAUNCH(2); else LAUNCH(3);
      (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
      float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
      if (r >= 2) t += ms;
    }
    return t / reps * 1e3f;
  };
  // an empty launch for the event overhead
  const float t3 = run(3), t2 = run(2), t1 = run(1);
  printf("variants (us, events incl. launch): no tile loads %.2f | no output stores %.2f | constant taps %.2f\n", t3, t2, t1);
  tot = run(0) * reps / 1e3f;
  if (hipGetLastError() != hipSuccess) { printf("launch failed\n"); return 1; }
  (void)hipMemcpy(hy.data(), dy, n * 2, hipMemcpyDeviceToHost);
  double worst = 0; int bad = 0;
  for (int s = 0; s < 4000; ++s) {
    const int b = rand() % B, c = rand() % C, t = rand() % Tp;
    double ref = 0;
    if (t < T) for (int j = 0; j < k; ++j) { const int ti = t + j - k / 2; if (ti >= 0 && ti < T) ref += (double)hw[(size_t)c * k + j] * bf2f(hx[((size_t)b * C + c) * Tp + ti]); }
    const double got = bf2f(hy[((size_t)b * C + c) * Tp + t]);
    const double err = fabs(got - ref);
    if (err > 0.02 + 0.01 * fabs(ref)) ++bad;
    worst = err > worst ? err : worst;
  }
  printf("time-major depthwise forward, B=%d C=%d T'=%d k=%d: %.2f us per launch (events, incl. launch); %d / 4000 samples off, worst abs err %.4f\n",
         B, C, T, k, tot / reps * 1e3, bad, worst);
  return bad != 0;
}
