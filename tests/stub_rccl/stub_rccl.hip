// TEST INFRASTRUCTURE - not part of the product.  A stand-in for librccl.so (selected with LASR_RCCL_PATH) that implements the
// eight entry points lightning_asr_amd/csrc/comm.hip binds, for SEVERAL PROCESSES SHARING ONE GPU: real RCCL refuses two ranks
// on one device, and the test box has one MI355X, so without this the library's own communicator (lasr_comm_*) would never see
// world > 1 before the driver's 8-GPU run.  Ranks meet in a POSIX shared-memory segment named after the unique id, exchange
// hipIpc handles of one staging buffer each, and every collective is
//     stage my operand -> barrier -> out = sum over ranks in rank order (same bits on every rank) -> barrier
// with the barriers as stream-ordered host callbacks (hipLaunchHostFunc) that spin on counters in the shared segment.
// Nothing here is tuned or meant to be: it checks ordering, bucket ranges, broadcast roots and the 1/world bookkeeping.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <atomic>
#include <fcntl.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <time.h>
#include <unistd.h>

namespace {

constexpr int kMaxRanks = 8;
constexpr size_t kStageBytes = 64u << 20;

struct Shared {
  std::atomic<int> ready[kMaxRanks];
  std::atomic<long long> arrive[kMaxRanks];    // per-rank count of barriers entered
  hipIpcMemHandle_t handle[kMaxRanks];
};

struct StubComm {
  int world = 1, rank = 0;
  char name[64];
  Shared* sh = nullptr;
  float* stage[kMaxRanks] = {};                // [rank] = my own allocation, peers through hipIpcOpenMemHandle
  long long executed = 0;                      // barriers EXECUTED so far by this rank (counted inside the callback, so a barrier
                                               // replayed from a captured hipGraph gets a fresh number each time it runs)
};

void barrier_cb(void* p) {
  StubComm* c = static_cast<StubComm*>(p);
  const long long target = ++c->executed;      // callbacks of one stream run in order: the k-th barrier of every rank pairs up
  c->sh->arrive[c->rank].store(target, std::memory_order_release);
  const time_t t0 = time(nullptr);
  for (int r = 0; r < c->world; ++r) {
    while (c->sh->arrive[r].load(std::memory_order_acquire) < target) {
      if (time(nullptr) - t0 > 120) { fprintf(stderr, "stub_rccl: rank %d timed out waiting for rank %d at barrier %lld\n", c->rank, r, target); abort(); }
      usleep(20);
    }
  }
}

hipError_t stream_barrier(StubComm* c, hipStream_t st) {
  if (c->world == 1) return hipSuccess;
  return hipLaunchHostFunc(st, barrier_cb, c);
}

struct Ptrs { const float* p[kMaxRanks]; };

__global__ void sum_kernel(Ptrs src, int world, float* out, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    float s = src.p[0][i];
    for (int r = 1; r < world; ++r) s += src.p[r][i];
    out[i] = s;
  }
}

ncclResult_t hip2nccl(hipError_t e) { return e == hipSuccess ? ncclSuccess : ncclUnhandledCudaError; }

// ---- CU-holding mode (what does sharing the CUs with RCCL's channel kernels cost the backward?) -------------------------
// Real RCCL keeps one persistent workgroup per channel resident for as long as a bucket is on the wire.  With
//   LASR_STUB_HOLD_CUS=n        every all-reduce is accompanied, on the same stream, by a kernel of n workgroups of 256 threads that
//                               each claim LASR_STUB_HOLD_LDS_KB of LDS (default 96: more than half a CU's 160 KB, so the n workgroups
//                               sit on n DIFFERENT CUs and no 144 KB GEMM / depthwise workgroup fits beside one - the pessimistic
//                               model of a channel kernel) and spin on the constant-rate clock for the time the
//   LASR_STUB_WIRE_GBS=g        payload would spend on the wire at algorithm bandwidth g GB/s (default 85: ~150 GB/s bus bandwidth on
//   LASR_STUB_LAT_US=l          a ring of 8) plus l microseconds of latency (default 20)
// - also on a 1-rank communicator, so bench.py's LASR_FORCE_OVERLAP=1 step on ONE GPU prices the interference for n = 8 .. 64.
// Every wave leaves the loop when the clock says so: the grid always drains.
int env_int(const char* k, int dflt) { const char* v = getenv(k); return v ? atoi(v) : dflt; }
double env_dbl(const char* k, double dflt) { const char* v = getenv(k); return v ? atof(v) : dflt; }

__global__ __launch_bounds__(256) void hold_kernel(long long ticks, float* sink) {
  extern __shared__ float lds[];
  const long long t0 = wall_clock64();               // constant 100 MHz counter
  float acc = 0.f;
  lds[threadIdx.x] = (float)threadIdx.x;
  while (wall_clock64() - t0 < ticks) {
    acc += lds[(threadIdx.x * 7 + (int)acc) & 255];  // a little LDS + VALU traffic, like a copy loop between its network waits
    __builtin_amdgcn_s_sleep(8);
  }
  if (acc == -1.f) sink[0] = acc;                    // (never true: keeps the loop)
}

hipError_t hold_cus(size_t bytes, hipStream_t st) {
  static const int n = env_int("LASR_STUB_HOLD_CUS", 0);
  if (n <= 0) return hipSuccess;
  static const double gbs = env_dbl("LASR_STUB_WIRE_GBS", 85.0), lat = env_dbl("LASR_STUB_LAT_US", 20.0);
  static const int lds_kb = env_int("LASR_STUB_HOLD_LDS_KB", 96);
  static float* sink = nullptr;
  if (!sink && hipMalloc(&sink, 256) != hipSuccess) return hipErrorOutOfMemory;
  const double us = lat + (double)bytes / (gbs * 1e3);
  static bool attr = false;
  if (!attr) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(hold_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr = true; }
  hipLaunchKernelGGL(hold_kernel, dim3(n), dim3(256), (size_t)lds_kb * 1024, st, (long long)(us * 100.0), sink);
  return hipGetLastError();
}

}  // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
  memset(id, 0, sizeof(*id));
  unsigned long long r[2] = {(unsigned long long)getpid() * 0x9E3779B97F4A7C15ull ^ (unsigned long long)time(nullptr), 0};
  FILE* f = fopen("/dev/urandom", "rb");
  if (f) { if (fread(r, sizeof(r), 1, f) != 1) r[1] = 1; fclose(f); }
  snprintf(id->internal, sizeof(id->internal), "/lasr_stub_%016llx%016llx", r[0], r[1]);
  return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t* out, int nranks, ncclUniqueId id, int rank) {
  if (nranks < 1 || nranks > kMaxRanks || rank < 0 || rank >= nranks) return ncclInvalidArgument;
  StubComm* c = new StubComm();
  c->world = nranks; c->rank = rank;
  strncpy(c->name, id.internal, sizeof(c->name) - 1);
  c->name[sizeof(c->name) - 1] = 0;
  const int fd = shm_open(c->name, O_CREAT | O_RDWR, 0600);
  if (fd < 0 || ftruncate(fd, sizeof(Shared)) != 0) return ncclSystemError;
  c->sh = static_cast<Shared*>(mmap(nullptr, sizeof(Shared), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0));
  close(fd);
  if (c->sh == MAP_FAILED) return ncclSystemError;
  if (hipMalloc(&c->stage[rank], kStageBytes) != hipSuccess) return ncclUnhandledCudaError;
  if (nranks > 1) {
    if (hipIpcGetMemHandle(&c->sh->handle[rank], c->stage[rank]) != hipSuccess) return ncclUnhandledCudaError;
    c->sh->ready[rank].store(1, std::memory_order_release);
    const time_t t0 = time(nullptr);
    for (int r = 0; r < nranks; ++r) {
      while (!c->sh->ready[r].load(std::memory_order_acquire)) {
        if (time(nullptr) - t0 > 120) return ncclSystemError;
        usleep(100);
      }
      if (r != rank && hipIpcOpenMemHandle(reinterpret_cast<void**>(&c->stage[r]), c->sh->handle[r], hipIpcMemLazyEnablePeerAccess) != hipSuccess)
        return ncclUnhandledCudaError;
    }
  }
  *out = reinterpret_cast<ncclComm_t>(c);
  return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm) {
  StubComm* c = reinterpret_cast<StubComm*>(comm);
  if (!c) return ncclSuccess;
  (void)hipDeviceSynchronize();
  for (int r = 0; r < c->world; ++r)
    if (r != c->rank && c->stage[r]) (void)hipIpcCloseMemHandle(c->stage[r]);
  // (my own staging buffer stays allocated until the process ends: a peer may still have it mapped)
  if (c->rank == 0) shm_unlink(c->name);
  munmap(c->sh, sizeof(Shared));
  delete c;
  return ncclSuccess;
}

ncclResult_t ncclAllReduce(const void* send, void* recv, size_t count, ncclDataType_t dt, ncclRedOp_t op, ncclComm_t comm, hipStream_t st) {
  StubComm* c = reinterpret_cast<StubComm*>(comm);
  if (!c || dt != ncclFloat32 || op != ncclSum) return ncclInvalidArgument;
  if (c->world == 1) {
    if (send != recv) { hipError_t e = hipMemcpyAsync(recv, send, count * sizeof(float), hipMemcpyDeviceToDevice, st); if (e != hipSuccess) return hip2nccl(e); }
    return hip2nccl(hold_cus(count * sizeof(float), st));
  }
  { hipError_t e = hold_cus(count * sizeof(float), st); if (e != hipSuccess) return hip2nccl(e); }
  const size_t chunk = kStageBytes / sizeof(float);
  for (size_t off = 0; off < count; off += chunk) {
    const size_t n = count - off < chunk ? count - off : chunk;
    hipError_t e = hipMemcpyAsync(c->stage[c->rank], static_cast<const float*>(send) + off, n * sizeof(float), hipMemcpyDeviceToDevice, st);
    if (e == hipSuccess) e = stream_barrier(c, st);                      // every rank has staged its operand
    if (e != hipSuccess) return hip2nccl(e);
    Ptrs p;
    for (int r = 0; r < kMaxRanks; ++r) p.p[r] = r < c->world ? c->stage[r] : nullptr;
    hipLaunchKernelGGL(sum_kernel, dim3(512), dim3(256), 0, st, p, c->world, static_cast<float*>(recv) + off, n);
    e = hipGetLastError();
    if (e == hipSuccess) e = stream_barrier(c, st);                      // every rank has read every staging buffer
    if (e != hipSuccess) return hip2nccl(e);
  }
  return ncclSuccess;
}

ncclResult_t ncclBroadcast(const void* send, void* recv, size_t count, ncclDataType_t dt, int root, ncclComm_t comm, hipStream_t st) {
  StubComm* c = reinterpret_cast<StubComm*>(comm);
  if (!c || dt != ncclFloat32 || root < 0 || root >= c->world) return ncclInvalidArgument;
  if (c->world == 1) return ncclSuccess;
  const size_t chunk = kStageBytes / sizeof(float);
  for (size_t off = 0; off < count; off += chunk) {
    const size_t n = count - off < chunk ? count - off : chunk;
    hipError_t e = hipSuccess;
    if (c->rank == root) e = hipMemcpyAsync(c->stage[root], static_cast<const float*>(send) + off, n * sizeof(float), hipMemcpyDeviceToDevice, st);
    if (e == hipSuccess) e = stream_barrier(c, st);
    if (e == hipSuccess && c->rank != root)
      e = hipMemcpyAsync(static_cast<float*>(recv) + off, c->stage[root], n * sizeof(float), hipMemcpyDeviceToDevice, st);
    if (e == hipSuccess) e = stream_barrier(c, st);
    if (e != hipSuccess) return hip2nccl(e);
  }
  return ncclSuccess;
}

ncclResult_t ncclGroupStart() { return ncclSuccess; }
ncclResult_t ncclGroupEnd() { return ncclSuccess; }
const char* ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "success" : "stub_rccl error"; }

}  // extern "C"
