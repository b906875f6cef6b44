// Dev microbenchmark: cost of accumulating per-column sums with device-scope f64 atomics at the END of a streaming kernel,
// the way a GEMM epilogue / BN-statistics pass would (W workgroups each add NCOL values into the same NCOL addresses),
// against writing per-workgroup partial rows.   hipcc --offload-arch=gfx950 -O3 -o build/atomic_f64 tools/micro/atomic_f64.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

// mode 0: partial rows (plain stores)   mode 1: f64 atomicAdd   mode 2: f32 atomicAdd   mode 3: nothing (stream only)
template <int MODE>
__global__ __launch_bounds__(256) void k(const uint4* __restrict__ p, size_t n16_per_wg, int ncol, double* sums, float* sums32, float* partials) {
  float acc = 0.f;
  const uint4* q = p + (size_t)blockIdx.x * n16_per_wg;
  for (size_t i = threadIdx.x; i < n16_per_wg; i += 256 * 4) {
    uint4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) { const size_t j = i + u * 256; v[u] = q[j < n16_per_wg ? j : n16_per_wg - 1]; }
#pragma unroll
    for (int u = 0; u < 4; ++u) acc += __uint_as_float(v[u].x) + __uint_as_float(v[u].w);
  }
  for (int c = threadIdx.x; c < ncol; c += 256) {
    const float val = acc * 1e-30f + (float)c;
    if (MODE == 0) partials[(size_t)blockIdx.x * ncol + c] = val;
    if (MODE == 1) atomicAdd(sums + c, (double)val);
    if (MODE == 2) atomicAdd(sums32 + c, val);
    if (MODE == 3 && val == -1.f) partials[c] = val;
  }
}

int main() {
  const size_t bytes = (size_t)64 << 20;
  uint4* buf; double* sums; float* sums32; float* partials;
  (void)hipMalloc(&buf, bytes); (void)hipMalloc(&sums, 8 * 4096); (void)hipMalloc(&sums32, 4 * 4096); (void)hipMalloc(&partials, (size_t)4096 * 4096 * 4);
  (void)hipMemset(buf, 0, bytes); (void)hipMemset(sums, 0, 8 * 4096); (void)hipMemset(sums32, 0, 4 * 4096);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int wgs[] = {252, 504, 1002};
  const int ncols[] = {512, 1024, 2048};
  for (int W : wgs)
    for (int ncol : ncols) {
      const size_t n16 = bytes / 16 / W;
      float t[4] = {0, 0, 0, 0};
      for (int mode = 0; mode < 4; ++mode) {
        for (int r = 0; r < 7; ++r) {
          (void)hipEventRecord(e0);
          if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(W), dim3(256), 0, 0, buf, n16, ncol, sums, sums32, partials);
          if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(W), dim3(256), 0, 0, buf, n16, ncol, sums, sums32, partials);
          if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(W), dim3(256), 0, 0, buf, n16, ncol, sums, sums32, partials);
          if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(W), dim3(256), 0, 0, buf, n16, ncol, sums, sums32, partials);
          (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
          float ms; (void)hipEventElapsedTime(&ms, e0, e1);
          if (r >= 2) t[mode] += ms / 5;
        }
      }
      printf("W=%4d ncol=%4d  stream-only %.1f us | partial rows %.1f | f64 atomics %.1f | f32 atomics %.1f\n", W, ncol, t[3] * 1e3, t[0] * 1e3,
             t[1] * 1e3, t[2] * 1e3);
    }
  double h[4]; (void)hipMemcpy(h, sums, 32, hipMemcpyDeviceToHost);
  printf("check %.1f %.1f\n", h[0], h[1]);
  return 0;
}
