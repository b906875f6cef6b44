// Dev microbenchmark: how fast the memory system takes 33 MB of bf16 GEMM-output stores under different
// workgroup -> address patterns (252 workgroups x 512 threads, each writing a 256 x 256 bf16 tile of two
// [16128][512] matrices).   hipcc --offload-arch=gfx950 -O3 -o build/store_pattern tools/micro/store_pattern.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>

__device__ __forceinline__ int xcd_remap(int orig, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = orig & 7;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
}

// pattern: 0 gemm epilogue (wave w: rows w*32 + it*2 + {0,1}, 512 B per row)   1 = 0 with the XCD remap
//          2 linear (workgroup t writes bytes [t*128K, (t+1)*128K))            3 = 2 with the XCD remap
//          4 rows it-major (row = it*16 + w*2 + {0,1})                           5 tile = 128 rows x full 1 KB rows
//          6 = 0 but each wave waits for its stores one by one (vmcnt(0) after each)
//          7 = 0 with 4 stores then a pause of ~1 us (staggered issue)
__global__ __launch_bounds__(512, 1) void store_kernel(char* C, int pattern, int pre_spin) {
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  int t = blockIdx.x;
  if (pattern == 1 || pattern == 3) t = xcd_remap(t, gridDim.x);
  const int p = t / 126, tt = t % 126, tm = tt >> 1, tn = tt & 1;
  char* base = C + (size_t)p * 16128 * 1024;
  uint4 v = make_uint4(tid, t, 3, 4);
  if (pre_spin) {   // emulate a K loop: every workgroup arrives at its stores at the same time anyway
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < (unsigned long long)pre_spin) {}
  }
  if (pattern == 2 || pattern == 3) {
#pragma unroll
    for (int it = 0; it < 16; ++it) *reinterpret_cast<uint4*>(C + (size_t)t * 131072 + (size_t)(it * 512 + tid) * 16) = v;
  } else if (pattern == 5) {
    // 128 rows x 1 KB: wave w rows w*16 + it (one full row per instruction)
    const int tm5 = t % 126;
#pragma unroll
    for (int it = 0; it < 16; ++it) {
      const int row = tm5 * 128 + wid * 16 + it;
      *reinterpret_cast<uint4*>(base + (size_t)row * 1024 + lane * 16) = v;
    }
  } else {
#pragma unroll
    for (int it = 0; it < 16; ++it) {
      const int lr = (pattern == 4) ? it * 16 + wid * 2 + (lane >> 5) : wid * 32 + it * 2 + (lane >> 5);
      const int row = tm * 256 + lr;
      *reinterpret_cast<uint4*>(base + (size_t)row * 1024 + tn * 512 + (lane & 31) * 16) = v;
      if (pattern == 6) __builtin_amdgcn_s_waitcnt(0);
      if (pattern == 7 && (it & 3) == 3) { const unsigned long long t0 = wall_clock64(); while (wall_clock64() - t0 < 100ull) {} }
    }
  }
}

int main() {
  char* C;
  const size_t bytes = (size_t)2 * 16128 * 1024;
  hipMalloc(&C, bytes);
  hipMemset(C, 0, bytes);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int spin = 0; spin <= 1; ++spin)
    for (int pat = 0; pat < 8; ++pat) {
      for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(store_kernel, dim3(252), dim3(512), 0, 0, C, pat, spin * 500);
      hipDeviceSynchronize();
      const int reps = 20;
      hipEventRecord(e0, 0);
      for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(store_kernel, dim3(252), dim3(512), 0, 0, C, pat, spin * 500);
      hipEventRecord(e1, 0);
      hipEventSynchronize(e1);
      float ms = 0;
      hipEventElapsedTime(&ms, e0, e1);
      printf("pattern %d pre_spin %d us: %.1f us per launch (%.2f TB/s)\n", pat, spin * 5, ms * 1e3 / reps, bytes / (ms * 1e-3 / reps) / 1e12);
    }
  return 0;
}
