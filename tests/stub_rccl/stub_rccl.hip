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
    if (send != recv) return hip2nccl(hipMemcpyAsync(recv, send, count * sizeof(float), hipMemcpyDeviceToDevice, st));
    return ncclSuccess;
  }
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
