// Dev microbenchmark: cost of straight-line (executed-once) code.  Each workgroup runs N dependent-free FMAs, fully
// unrolled (code size ~ 8 N bytes), once; compared with the same N FMAs as a rolled loop of 64.
// hipcc --offload-arch=gfx950 -O3 -o build/icache tools/micro/icache.hip
#include <hip/hip_runtime.h>
#include <stdio.h>

template <int N, bool ROLLED>
__global__ __launch_bounds__(512, 1) void code_kernel(float* out, float a, float b) {
  float x[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) x[i] = threadIdx.x * 0.001f + i;
  if (ROLLED) {
#pragma unroll 1
    for (int j = 0; j < N / 64; ++j) {
#pragma unroll
      for (int i = 0; i < 64; ++i) x[i & 7] = __builtin_fmaf(x[i & 7], a, b + (float)i);
    }
  } else {
#pragma unroll
    for (int i = 0; i < N; ++i) x[i & 7] = __builtin_fmaf(x[i & 7], a, b + (float)(i & 63));
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += x[i];
  if (s == 12345.678f) out[0] = s;
}

template <int N, bool ROLLED>
void run(float* out, const char* what) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((code_kernel<N, ROLLED>), dim3(256), dim3(512), 0, 0, out, 1.0001f, 0.5f);
  hipDeviceSynchronize();
  const int reps = 20;
  hipEventRecord(e0, 0);
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((code_kernel<N, ROLLED>), dim3(256), dim3(512), 0, 0, out, 1.0001f, 0.5f);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  printf("%6d FMAs per thread, %s: %.1f us per launch\n", N, what, ms * 1e3 / reps);
}

int main() {
  float* out;
  hipMalloc(&out, 1024);
  run<256, false>(out, "unrolled"); run<256, true>(out, "rolled  ");
  run<1024, false>(out, "unrolled"); run<1024, true>(out, "rolled  ");
  run<2048, false>(out, "unrolled"); run<2048, true>(out, "rolled  ");
  run<4096, false>(out, "unrolled"); run<4096, true>(out, "rolled  ");
  run<8192, false>(out, "unrolled"); run<8192, true>(out, "rolled  ");
  return 0;
}
