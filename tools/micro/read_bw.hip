// Dev microbenchmark: streaming-read bandwidth of a 16 - 96 MB buffer, warm (just read) and cold (768 MB of other
// traffic in between evicts the 256 MB memory-side cache).  hipcc --offload-arch=gfx950 -O3 -o build/read_bw tools/micro/read_bw.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

template <int INFLIGHT>
__global__ __launch_bounds__(256) void read_kernel(const uint4* __restrict__ p, size_t n16, uint32_t* out) {
  uint32_t acc = 0;
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += stride * INFLIGHT) {
    uint4 v[INFLIGHT];
#pragma unroll
    for (int u = 0; u < INFLIGHT; ++u) { const size_t j = i + u * stride; v[u] = p[j < n16 ? j : n16 - 1]; }
#pragma unroll
    for (int u = 0; u < INFLIGHT; ++u) acc ^= v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
  }
  if (acc == 0x12345678u) out[0] = acc;
}
__global__ void scrub_kernel(uint4* p, size_t n16) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) { uint4 v = p[i]; v.x += 1; p[i] = v; }
}

int main() {
  const size_t scrub_bytes = (size_t)768 << 20;
  uint4 *buf, *scrub; uint32_t* out;
  (void)hipMalloc(&buf, (size_t)96 << 20); (void)hipMalloc(&scrub, scrub_bytes); (void)hipMalloc(&out, 64);
  (void)hipMemset(buf, 1, (size_t)96 << 20); (void)hipMemset(scrub, 1, scrub_bytes);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int mbs[] = {49, 96};
  const int grids[] = {512, 2048, 8192};
  const int infl[] = {4, 8, 16};
  for (int mb : mbs)
    for (int grid : grids)
     for (int inflight : infl)
      for (int cold = 1; cold <= 1; ++cold) {
        const size_t n16 = ((size_t)mb << 20) / 16;
        float tot = 0;
        const int reps = 6;
        for (int r = 0; r < reps + 1; ++r) {
          if (cold) hipLaunchKernelGGL(scrub_kernel, dim3(4096), dim3(256), 0, 0, scrub, scrub_bytes / 16);
          (void)hipEventRecord(e0, 0);
          if (inflight == 4) hipLaunchKernelGGL(read_kernel<4>, dim3(grid), dim3(256), 0, 0, buf, n16, out);
          else if (inflight == 8) hipLaunchKernelGGL(read_kernel<8>, dim3(grid), dim3(256), 0, 0, buf, n16, out);
          else hipLaunchKernelGGL(read_kernel<16>, dim3(grid), dim3(256), 0, 0, buf, n16, out);
          (void)hipEventRecord(e1, 0);
          (void)hipEventSynchronize(e1);
          float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
          if (r > 0) tot += ms;
        }
        const float us = tot / reps * 1e3f;
        printf("%3d MB, %4d workgroups x %2d loads in flight, %s: %6.1f us (%.2f TB/s incl. launch)\n", mb, grid, inflight, cold ? "cold" : "warm", us, mb * 1.048576f / us);
      }
  return 0;
}
