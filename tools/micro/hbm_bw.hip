// Dev microbenchmark (round 4): what this box's memory system delivers to plain streaming kernels - the yardstick DESIGN 4
// prices the BN / depthwise passes against.  Replaces the round-3 reading of read_bw.hip, whose "cold" case dirtied 768 MB
// right before every timed read (reads fighting a write-back) and timed 49-96 MB buffers with the launch included.
//   hipcc --offload-arch=gfx950 -O3 -o build/hbm_bw tools/micro/hbm_bw.hip && build/hbm_bw
// Cases:
//   A  read 1.2 GB, nothing else going on                      (the guide's 6.0-6.3 TB/s case)
//   B  read 49 / 96 MB: warm (just read), after a CLEAN eviction (1.5 GB of reads in between), after a DIRTY one (1.5 GB written)
//   C  write 1.2 GB; copy 0.6 -> 0.6 GB
//   D  the saved-for-backward pattern: write 96 MB (default / non-temporal stores), 1.5 GB of other read+write traffic, read it back
//      (default / non-temporal loads): does keeping write-once-read-much-later tensors out of the memory-side cache help either side?
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int INFLIGHT, bool NT>
__global__ __launch_bounds__(256) void read_kernel(const u32x4* __restrict__ p, size_t n16, uint32_t* out) {
  uint32_t acc = 0;
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += stride * INFLIGHT) {
    u32x4 v[INFLIGHT];
#pragma unroll
    for (int u = 0; u < INFLIGHT; ++u) {
      const size_t j = i + u * stride;
      const u32x4* q = p + (j < n16 ? j : n16 - 1);
      v[u] = NT ? __builtin_nontemporal_load(q) : *q;
    }
#pragma unroll
    for (int u = 0; u < INFLIGHT; ++u) acc ^= v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
  }
  if (acc == 0x12345678u) out[0] = acc;
}

template <bool NT>
__global__ __launch_bounds__(256) void write_kernel(u32x4* __restrict__ p, size_t n16, uint32_t seed) {
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += stride) {
    u32x4 v = {seed, (uint32_t)i, seed ^ 7u, 1u};
    if (NT) __builtin_nontemporal_store(v, p + i); else p[i] = v;
  }
}

__global__ __launch_bounds__(256) void copy_kernel(const u32x4* __restrict__ src, u32x4* __restrict__ dst, size_t n16) {
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += stride * 4) {
    u32x4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) { const size_t j = i + u * stride; v[u] = src[j < n16 ? j : n16 - 1]; }
#pragma unroll
    for (int u = 0; u < 4; ++u) { const size_t j = i + u * stride; if (j < n16) dst[j] = v[u]; }
  }
}

static hipEvent_t e0, e1;
template <typename F>
static float timed_us(F&& f) {
  (void)hipEventRecord(e0, 0);
  f();
  (void)hipEventRecord(e1, 0);
  (void)hipEventSynchronize(e1);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e3f;
}

int main() {
  const size_t big = (size_t)1200 << 20, evict = (size_t)1536 << 20, small_max = (size_t)96 << 20;
  u32x4 *A, *E, *S;
  uint32_t* out;
  if (hipMalloc(&A, big) != hipSuccess || hipMalloc(&E, evict) != hipSuccess || hipMalloc(&S, small_max) != hipSuccess || hipMalloc(&out, 64) != hipSuccess) {
    printf("alloc failed\n");
    return 1;
  }
  (void)hipMemset(A, 1, big); (void)hipMemset(E, 2, evict); (void)hipMemset(S, 3, small_max);
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  (void)hipDeviceSynchronize();
  const int reps = 8;
  auto rd = [&](const u32x4* p, size_t bytes, int grid, bool nt) {
    if (nt) hipLaunchKernelGGL((read_kernel<8, true>), dim3(grid), dim3(256), 0, 0, p, bytes / 16, out);
    else hipLaunchKernelGGL((read_kernel<8, false>), dim3(grid), dim3(256), 0, 0, p, bytes / 16, out);
  };
  auto wr = [&](u32x4* p, size_t bytes, int grid, bool nt, uint32_t seed) {
    if (nt) hipLaunchKernelGGL((write_kernel<true>), dim3(grid), dim3(256), 0, 0, p, bytes / 16, seed);
    else hipLaunchKernelGGL((write_kernel<false>), dim3(grid), dim3(256), 0, 0, p, bytes / 16, seed);
  };
  // ---- A: 1.2 GB reads
  for (int grid : {2048, 8192, 32768}) {
    float tot = 0;
    for (int r = 0; r <= reps; ++r) { const float us = timed_us([&] { rd(A, big, grid, false); }); if (r) tot += us; }
    printf("A  read 1200 MB, %5d workgroups x 8 loads in flight: %7.1f us = %.2f TB/s\n", grid, tot / reps, (double)big / (tot / reps) / 1e6);
  }
  {
    float tot = 0;
    for (int r = 0; r <= reps; ++r) { const float us = timed_us([&] { rd(A, big, 8192, true); }); if (r) tot += us; }
    printf("A  read 1200 MB, non-temporal loads                  : %7.1f us = %.2f TB/s\n", tot / reps, (double)big / (tot / reps) / 1e6);
  }
  // ---- C: writes / copy
  for (int nt = 0; nt < 2; ++nt) {
    float tot = 0;
    for (int r = 0; r <= reps; ++r) { const float us = timed_us([&] { wr(A, big, 8192, nt, r); }); if (r) tot += us; }
    printf("C  write 1200 MB (%s stores): %7.1f us = %.2f TB/s\n", nt ? "non-temporal" : "default", tot / reps, (double)big / (tot / reps) / 1e6);
  }
  {
    float tot = 0;
    for (int r = 0; r <= reps; ++r) {
      const float us = timed_us([&] { hipLaunchKernelGGL(copy_kernel, dim3(8192), dim3(256), 0, 0, A, A + big / 32, big / 32); });
      if (r) tot += us;
    }
    printf("C  copy 600 -> 600 MB: %7.1f us = %.2f TB/s (read + write)\n", tot / reps, (double)big / (tot / reps) / 1e6);
  }
  // ---- B: small buffers, three cache states
  for (size_t mb : {(size_t)49, (size_t)96}) {
    const size_t bytes = mb << 20;
    for (int state = 0; state < 3; ++state) {
      float tot = 0;
      for (int r = 0; r <= reps; ++r) {
        if (state == 0) rd(S, bytes, 2048, false);
        if (state == 1) rd(E, evict, 8192, false);
        if (state == 2) wr(E, evict, 8192, false, r);
        const float us = timed_us([&] { rd(S, bytes, 2048, false); });
        if (r) tot += us;
      }
      const char* names[] = {"warm (just read)", "after 1.5 GB of reads (clean eviction)", "after 1.5 GB of writes (dirty lines draining)"};
      printf("B  read %3zu MB, %-46s: %6.1f us = %.2f TB/s (launch included)\n", mb, names[state], tot / reps, (double)bytes / (tot / reps) / 1e6);
    }
  }
  // ---- D: write once, read much later
  for (int ntw = 0; ntw < 2; ++ntw)
    for (int ntr = 0; ntr < 2; ++ntr) {
      float tw = 0, tr = 0, tmid = 0;
      const size_t bytes = (size_t)96 << 20;
      for (int r = 0; r <= reps; ++r) {
        rd(E, evict, 8192, false);                                   // settle
        const float a = timed_us([&] { wr(S, bytes, 2048, ntw, r); });
        const float m = timed_us([&] { hipLaunchKernelGGL(copy_kernel, dim3(8192), dim3(256), 0, 0, E, E + evict / 32, evict / 32); });
        const float b = timed_us([&] { rd(S, bytes, 2048, ntr); });
        if (r) { tw += a; tmid += m; tr += b; }
      }
      printf("D  96 MB written with %-12s stores: %5.1f us; 1.5 GB of traffic in between: %6.1f us; read back with %-12s loads: %5.1f us\n",
             ntw ? "non-temporal" : "default", tw / reps, tmid / reps, ntr ? "non-temporal" : "default", tr / reps);
    }
  return 0;
}
