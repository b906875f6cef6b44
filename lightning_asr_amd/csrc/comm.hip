// Data-parallel gradient exchange: RCCL called directly from the library (lasr_comm_init / allreduce / destroy).
// Replaces what the reference gets implicitly from Lightning's DDP plugin (conf/conf.yaml:30 `accelerator: ddp`,
// train.py:239): a SUM all-reduce of the gradients over NCCL, overlapped with backward.
//
// One communicator per process (one process per GPU).  The collectives run on a LIBRARY-OWNED side stream, ordered
// against the compute stream by events only:
//   lasr_comm_allreduce(comm, buf, n, producer)  event on `producer` (everything enqueued so far: the backward stage that
//                                                finalised this bucket) -> side stream waits -> ncclAllReduce in place
//   lasr_comm_wait(comm, consumer)               event on the side stream -> `consumer` waits (the optimiser launch)
// so RCCL's kernels overlap whatever the compute stream runs after the producing stage.  No host synchronisation.
// librccl.so is loaded on first use (dlopen): single-GPU users never pay for it, and the library has no link-time
// dependency on RCCL.  The 128-byte unique id is created on rank 0 and carried to the other ranks by the host's own
// rendez-vous (torch.distributed's store / any broadcast): that bootstrap is the only thing the host framework does.
#include "common.h"

#include <dlfcn.h>
#include <vector>
#include <rccl/rccl.h>

namespace lasr {

struct RcclApi {
  void* handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

static RcclApi g_rccl;
static constexpr int kDefaultMaxChannels = 0;   // 0 = RCCL's own choice (see lasr_comm_init)

static int load_rccl() {
  if (g_rccl.handle) return 0;
  const char* names[] = {getenv("LASR_RCCL_PATH"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
  void* h = nullptr;
  for (const char* n : names) {
    if (!n) continue;
    h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
    if (h) break;
  }
  if (!h) return fail(LASR_E_ARG, "lasr_comm: cannot load librccl.so (%s)", dlerror());
  RcclApi a;
  a.handle = h;
#define LASR_SYM(field, sym)                                                               \
  a.field = reinterpret_cast<decltype(a.field)>(dlsym(h, sym));                            \
  if (!a.field) return fail(LASR_E_ARG, "lasr_comm: librccl.so has no symbol %s", sym);
  LASR_SYM(GetUniqueId, "ncclGetUniqueId")
  LASR_SYM(CommInitRank, "ncclCommInitRank")
  LASR_SYM(CommDestroy, "ncclCommDestroy")
  LASR_SYM(AllReduce, "ncclAllReduce")
  LASR_SYM(Broadcast, "ncclBroadcast")
  LASR_SYM(GroupStart, "ncclGroupStart")
  LASR_SYM(GroupEnd, "ncclGroupEnd")
  LASR_SYM(GetErrorString, "ncclGetErrorString")
#undef LASR_SYM
  g_rccl = a;
  return 0;
}

static int nccl_fail(ncclResult_t r, const char* what) {
  return fail((int)r > 0 ? 1000 + (int)r : 999, "%s: %s", what, g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "rccl error");
}
#define LASR_NCCL(expr, what)                                  \
  do {                                                         \
    ncclResult_t r__ = (expr);                                 \
    if (r__ != ncclSuccess) return ::lasr::nccl_fail(r__, what); \
  } while (0)
#define LASR_HIP(expr, what)                                   \
  do {                                                         \
    hipError_t e__ = (expr);                                   \
    if (e__ != hipSuccess) return ::lasr::hip_fail(e__, what); \
  } while (0)

}  // namespace lasr

using namespace lasr;

// one timed collective (lasr_comm_timing): events on the side stream around the ncclAllReduce / group
struct CommRec { hipEvent_t a, b; double bytes; };
// one timed lasr_comm_wait: events on the CONSUMER stream right before and right after its wait for the side stream -
// their distance is the part of the exchange that backward did not hide (what the optimiser launch actually waited)
struct WaitRec { hipEvent_t a, b; };

struct lasr_comm {
  ncclComm_t comm = nullptr;
  hipStream_t side = nullptr;        // library-owned: every collective of this communicator runs here
  hipEvent_t ev_in = nullptr;        // producer stream -> side stream
  hipEvent_t ev_out = nullptr;       // side stream -> consumer stream
  int world = 1, rank = 0, device = 0;
  int64_t calls = 0;
  bool timing = false;
  std::vector<CommRec> recs;
  std::vector<WaitRec> waits;
  std::vector<hipEvent_t> pool;
  hipEvent_t timed_event() {
    if (!pool.empty()) { hipEvent_t e = pool.back(); pool.pop_back(); return e; }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
  }
};

extern "C" int lasr_comm_unique_id(void* id_out, size_t id_bytes) {
  LASR_CHECK_ARG(id_out && id_bytes >= LASR_COMM_ID_BYTES, "lasr_comm_unique_id: need a %d-byte buffer", LASR_COMM_ID_BYTES);
  static_assert(sizeof(ncclUniqueId) == LASR_COMM_ID_BYTES, "unique id size");
  LASR_TRY(load_rccl());
  ncclUniqueId id;
  LASR_NCCL(g_rccl.GetUniqueId(&id), "ncclGetUniqueId");
  memcpy(id_out, &id, sizeof(id));
  return 0;
}

extern "C" int lasr_comm_init(lasr_comm_t** out, const void* unique_id, size_t id_bytes, int world, int rank, int device) {
  LASR_CHECK_ARG(out && unique_id && id_bytes >= LASR_COMM_ID_BYTES, "lasr_comm_init: null pointer / short id");
  LASR_CHECK_ARG(world >= 1 && rank >= 0 && rank < world && device >= 0, "lasr_comm_init: world=%d rank=%d device=%d", world, rank, device);
  LASR_TRY(load_rccl());
  LASR_HIP(hipSetDevice(device), "hipSetDevice");
  // RCCL's channel count = the number of persistent workgroups its collective kernels keep resident while a bucket is on the wire.
  // Measured on one GPU with the stand-in's CU-holding mode (profiles/r04_cu_sharing.json): the backward loses the SAME
  // ~0.11 ms whether 2 or 64 CUs are held and whatever LDS they claim - what costs is a second busy hardware queue for the length of
  // the window (+0.09 / +0.11 / +0.15 ms for a 0.12 / 0.22 / 0.44 ms window), not the CUs taken.  So the exchange should be as SHORT
  // as RCCL can make it: no channel cap by default.  LASR_COMM_MAX_CHANNELS=n (> 0) sets NCCL_MAX_NCHANNELS for experiments on the
  // real fabric; an NCCL_MAX_NCHANNELS already in the environment wins.
  {
    const char* capv = getenv("LASR_COMM_MAX_CHANNELS");
    const int cap = capv ? atoi(capv) : kDefaultMaxChannels;
    if (cap > 0 && !getenv("NCCL_MAX_NCHANNELS")) {
      char buf[16];
      snprintf(buf, sizeof(buf), "%d", cap);
      setenv("NCCL_MAX_NCHANNELS", buf, 0);
    }
  }
  lasr_comm* c = new lasr_comm();
  c->world = world; c->rank = rank; c->device = device;
  ncclUniqueId id;
  memcpy(&id, unique_id, sizeof(id));
  ncclResult_t r = g_rccl.CommInitRank(&c->comm, world, id, rank);
  if (r != ncclSuccess) { delete c; return nccl_fail(r, "ncclCommInitRank"); }
  hipError_t e = hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_in, hipEventDisableTiming);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_out, hipEventDisableTiming);
  if (e != hipSuccess) { lasr_comm_destroy(c); return hip_fail(e, "lasr_comm_init: stream / events"); }
  *out = c;
  return 0;
}

extern "C" int lasr_comm_destroy(lasr_comm_t* c) {
  if (!c) return 0;
  if (c->side) (void)hipStreamSynchronize(c->side);
  if (c->comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(c->comm);
  if (c->ev_in) (void)hipEventDestroy(c->ev_in);
  if (c->ev_out) (void)hipEventDestroy(c->ev_out);
  for (auto& r : c->recs) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
  for (auto& w : c->waits) { (void)hipEventDestroy(w.a); (void)hipEventDestroy(w.b); }
  for (auto& e : c->pool) (void)hipEventDestroy(e);
  if (c->side) (void)hipStreamDestroy(c->side);
  delete c;
  return 0;
}

extern "C" int lasr_comm_world(const lasr_comm_t* c) { return c ? c->world : -1; }
extern "C" int lasr_comm_rank(const lasr_comm_t* c) { return c ? c->rank : -1; }

// the side stream picks up after everything enqueued on `producer` so far
static int order_after(lasr_comm* c, void* producer) {
  LASR_HIP(hipEventRecord(c->ev_in, as_stream(producer)), "hipEventRecord(producer)");
  LASR_HIP(hipStreamWaitEvent(c->side, c->ev_in, 0), "hipStreamWaitEvent(side)");
  return 0;
}

extern "C" int lasr_comm_allreduce(lasr_comm_t* c, float* buf, int64_t count, void* producer_stream) {
  LASR_CHECK_ARG(c && buf && count > 0, "lasr_comm_allreduce: null pointer / empty bucket");
  LASR_TRY(order_after(c, producer_stream));
  CommRec rec{nullptr, nullptr, (double)count * sizeof(float)};
  if (c->timing) { rec.a = c->timed_event(); rec.b = c->timed_event(); (void)hipEventRecord(rec.a, c->side); }
  LASR_NCCL(g_rccl.AllReduce(buf, buf, (size_t)count, ncclFloat32, ncclSum, c->comm, c->side), "ncclAllReduce");
  if (c->timing) { (void)hipEventRecord(rec.b, c->side); c->recs.push_back(rec); }
  c->calls += 1;
  return 0;
}

extern "C" int lasr_comm_allreduce_ranges(lasr_comm_t* c, float* base, const int64_t* lo, const int64_t* hi, int n_ranges,
                                          void* producer_stream) {
  LASR_CHECK_ARG(c && base && lo && hi && n_ranges > 0 && n_ranges <= 64, "lasr_comm_allreduce_ranges: bad argument");
  for (int i = 0; i < n_ranges; ++i) LASR_CHECK_ARG(lo[i] >= 0 && hi[i] > lo[i], "lasr_comm_allreduce_ranges: range %d", i);
  LASR_TRY(order_after(c, producer_stream));
  CommRec rec{nullptr, nullptr, 0.0};
  for (int i = 0; i < n_ranges; ++i) rec.bytes += (double)(hi[i] - lo[i]) * sizeof(float);
  if (c->timing) { rec.a = c->timed_event(); rec.b = c->timed_event(); (void)hipEventRecord(rec.a, c->side); }
  LASR_NCCL(g_rccl.GroupStart(), "ncclGroupStart");      // one fused launch for the pieces of a bucket
  for (int i = 0; i < n_ranges; ++i) {
    ncclResult_t r = g_rccl.AllReduce(base + lo[i], base + lo[i], (size_t)(hi[i] - lo[i]), ncclFloat32, ncclSum, c->comm, c->side);
    if (r != ncclSuccess) { (void)g_rccl.GroupEnd(); return nccl_fail(r, "ncclAllReduce"); }
  }
  LASR_NCCL(g_rccl.GroupEnd(), "ncclGroupEnd");
  if (c->timing) { (void)hipEventRecord(rec.b, c->side); c->recs.push_back(rec); }
  c->calls += 1;
  return 0;
}

extern "C" int lasr_comm_broadcast(lasr_comm_t* c, float* buf, int64_t count, int root, void* producer_stream) {
  LASR_CHECK_ARG(c && buf && count > 0 && root >= 0 && root < c->world, "lasr_comm_broadcast: bad argument");
  LASR_TRY(order_after(c, producer_stream));
  LASR_NCCL(g_rccl.Broadcast(buf, buf, (size_t)count, ncclFloat32, root, c->comm, c->side), "ncclBroadcast");
  return 0;
}

extern "C" int lasr_comm_wait(lasr_comm_t* c, void* consumer_stream) {
  LASR_CHECK_ARG(c, "lasr_comm_wait: null communicator");
  LASR_HIP(hipEventRecord(c->ev_out, c->side), "hipEventRecord(side)");
  WaitRec w{nullptr, nullptr};
  if (c->timing) { w.a = c->timed_event(); w.b = c->timed_event(); (void)hipEventRecord(w.a, as_stream(consumer_stream)); }
  LASR_HIP(hipStreamWaitEvent(as_stream(consumer_stream), c->ev_out, 0), "hipStreamWaitEvent(consumer)");
  if (c->timing) { (void)hipEventRecord(w.b, as_stream(consumer_stream)); c->waits.push_back(w); }
  return 0;
}

// Timing of the exchange for bench.py's `comm` record (eager steps only: the events are recorded by these host calls).
extern "C" int lasr_comm_timing(lasr_comm_t* c, int on) {
  LASR_CHECK_ARG(c, "lasr_comm_timing: null communicator");
  if (on) {
    for (auto& r : c->recs) { c->pool.push_back(r.a); c->pool.push_back(r.b); }
    for (auto& w : c->waits) { c->pool.push_back(w.a); c->pool.push_back(w.b); }
    c->recs.clear(); c->waits.clear();
  }
  c->timing = on != 0;
  return 0;
}

extern "C" int lasr_comm_timing_collect(lasr_comm_t* c, int max_recs, double* coll_us, double* coll_bytes, int* n_coll, double* wait_us,
                                        int* n_waits) {
  LASR_CHECK_ARG(c && n_coll && n_waits && max_recs >= 0, "lasr_comm_timing_collect: bad argument");
  int nc = 0, nw = 0;
  for (auto& r : c->recs) {
    hipError_t e = hipEventSynchronize(r.b);
    if (e != hipSuccess) return hip_fail(e, "lasr_comm_timing_collect");
    float t = 0.f;
    e = hipEventElapsedTime(&t, r.a, r.b);
    if (e != hipSuccess) return hip_fail(e, "lasr_comm_timing_collect");
    if (nc < max_recs && coll_us && coll_bytes) { coll_us[nc] = 1e3 * (double)t; coll_bytes[nc] = r.bytes; }
    ++nc;
  }
  for (auto& w : c->waits) {
    hipError_t e = hipEventSynchronize(w.b);
    if (e != hipSuccess) return hip_fail(e, "lasr_comm_timing_collect");
    float t = 0.f;
    e = hipEventElapsedTime(&t, w.a, w.b);
    if (e != hipSuccess) return hip_fail(e, "lasr_comm_timing_collect");
    if (nw < max_recs && wait_us) wait_us[nw] = 1e3 * (double)t;
    ++nw;
  }
  *n_coll = nc; *n_waits = nw;
  return 0;
}
